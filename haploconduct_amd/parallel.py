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

    Per batch: device stream compaction of the non-dropped records (`hc_compact_device`), one kernel that tags
    them with their global candidate index (`hc_pack_rows_device`, 32-byte rows), then two asynchronous all-gathers
    over RCCL: the counts and the rows padded to a fixed capacity.  All of it runs on a SIDE stream that only waits
    for the scoring kernel of its own batch, so with two result buffers the scoring kernel of batch i+1 starts
    right behind that of batch i and the collection of batch i overlaps it.  Nothing synchronises with the host;
    `collect()` is the host side: waits, checks the capacity, trims.

    Protocol per batch:  before_write(results) -> launch the scoring kernel into `results` on the current stream
    -> step(results).  Use at least two `results` tensors in turn; with one, before_write serialises.
    """

    def __init__(self, scorer, n_local, base_index, cap_rows, group=None, depth=2):
        self.sc, self.n, self.base, self.cap, self.group, self.depth = scorer, int(n_local), int(base_index), int(cap_rows), group, depth
        self.world = dist.get_world_size(group)
        dev = torch.device("cuda", torch.cuda.current_device())
        self.side = torch.cuda.Stream(device=dev)
        # zero-initialised: entries beyond the count of a batch are stale but always valid indices
        self.idx = torch.zeros(max(self.n, 1), dtype=torch.int32, device=dev)
        rows = self.cap + 1  # row 0 carries the count: one all-gather per batch is the whole all-gather-v
        self.bufs = [{"count": torch.zeros(1, dtype=torch.int64, device=dev),
                      "payload": torch.zeros((rows, 4), dtype=torch.int64, device=dev),
                      "all": torch.zeros((self.world * rows, 4), dtype=torch.int64, device=dev),
                      "scored": torch.cuda.Event(), "packed": torch.cuda.Event(), "work": None, "unordered": False}
                     for _ in range(depth)]
        self.i = 0
        self.packed = {}  # data_ptr of a results tensor -> event: its rows have been packed, it may be overwritten

    def before_write(self, d_results):
        """The current stream waits until the batch that last used `d_results` has been read out of it."""
        ev = self.packed.get(d_results.data_ptr())
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)

    def step(self, d_results):
        """d_results: uint8/int64 CUDA tensor holding this rank's n hc_result_rec, written by work already enqueued
        on the current stream.  Enqueues the collection on the side stream; returns the buffer set."""
        b = self.bufs[self.i % self.depth]
        self.i += 1
        b["unordered"] = False
        b["scored"].record(torch.cuda.current_stream())
        with torch.cuda.stream(self.side):
            self.side.wait_event(b["scored"])
            if b["work"] is not None:
                b["work"].wait()  # the batch that used these buffers `depth` batches ago has been gathered
            # (self.idx is shared: compaction and pack of consecutive batches are ordered on the side stream)
            self.sc.compact_pack_device(d_results.data_ptr(), self.n, self.idx.data_ptr(), b["count"].data_ptr(), self.cap, self.base,
                                        b["payload"].data_ptr(), self.side.cuda_stream)
            b["packed"].record(self.side)
            self.packed[d_results.data_ptr()] = b["packed"]
            b["work"] = dist.all_gather_into_tensor(b["all"], b["payload"], group=self.group, async_op=True)
        return b

    def score_step(self, d_in_ptr, d_results):
        """Scoring and collection of one batch with the payload written by the scoring kernel itself
        (`hc_score_pack_device`): no compaction pass.  d_in_ptr: device pointer of this rank's n candidate records;
        d_results: CUDA tensor receiving the n result records.  The all-gather runs on the side stream."""
        b = self.bufs[self.i % self.depth]
        self.i += 1
        main = torch.cuda.current_stream()
        if b["work"] is not None:
            b["work"].wait()  # the payload buffer of `depth` batches ago has been gathered
        b["unordered"] = self.sc.score_pack_device(d_in_ptr, self.n, d_results.data_ptr(), self.cap, self.base, b["payload"].data_ptr(),
                                                   main.cuda_stream)
        b["scored"].record(main)
        with torch.cuda.stream(self.side):
            self.side.wait_event(b["scored"])
            b["work"] = dist.all_gather_into_tensor(b["all"], b["payload"], group=self.group, async_op=True)
        return b

    def collect(self, b):
        """Host side of one batch: (rows [sum k_r, 4] int64 ordered by global index, counts per rank)."""
        with torch.cuda.stream(self.side):
            if b["work"] is not None:
                b["work"].wait()
        self.side.synchronize()
        rows = self.cap + 1
        counts = [int(b["all"][r * rows, 0]) for r in range(self.world)]
        if max(counts) > self.cap:
            raise OverflowError(f"a rank produced {max(counts)} records, capacity is {self.cap}: rerun the batch with a larger cap_rows")
        out = torch.cat([b["all"][r * rows + 1: r * rows + 1 + counts[r]] for r in range(self.world)], dim=0)
        if b.get("unordered"):  # rows appended by the scoring kernel arrive in any order; they carry their index
            out = out[torch.argsort(out[:, 0])]
        return out, counts

    def finish(self):
        with torch.cuda.stream(self.side):
            for b in self.bufs:
                if b["work"] is not None:
                    b["work"].wait()
                b["work"] = None
        self.side.synchronize()
        torch.cuda.current_stream().synchronize()
