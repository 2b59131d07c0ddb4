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


def pack_payload(results, base_index, cap_rows, shuffle_seed=None):
    """The collection payload of one batch, built on the host (numpy) — the layout `hc_score_pack_device` /
    `hc_compact_pack_device` write on the device: int64 rows [index, x1 bits, x2 bits, mm | n_cls << 32], row 0 =
    [count, 0, 0, 0], then the records whose class is not DROP.  shuffle_seed: emit the rows in a random order,
    as the fused scoring kernel does.  Used where no device is involved (host callers, the gloo tests)."""
    res = np.asarray(results)
    cls = res["n_cls"] >> 28
    kept = np.nonzero(cls != 0)[0]
    if shuffle_seed is not None:
        kept = kept[np.random.default_rng(shuffle_seed).permutation(kept.size)]
    out = np.zeros((cap_rows + 1, 4), np.int64)
    out[0, 0] = kept.size
    k = kept[:cap_rows]
    out[1:1 + k.size, 0] = base_index + k
    out[1:1 + k.size, 1] = res["x1"][k].view(np.int64)
    out[1:1 + k.size, 2] = res["x2"][k].view(np.int64)
    out[1:1 + k.size, 3] = res["mm"][k].astype(np.int64) | (res["n_cls"][k].astype(np.int64) << 32)
    return torch.from_numpy(out)


class _DoneWork:
    """Stand-in for an async work handle whose collective has already completed (the staged gloo path below)."""

    def wait(self):
        return True


class PayloadGather:
    """The all-gather-v of the collection payloads as ONE all-gather per batch, on any backend (nccl = RCCL, gloo):
    every rank contributes a fixed-capacity payload whose row 0 is its count; `depth` buffer sets in turn, the
    all-gather asynchronous; `collect` (host side) waits, checks the capacity, trims by the counts, and sorts the
    rows by global index when their producer wrote them in no particular order."""

    def __init__(self, cap_rows, group=None, depth=2, device=None):
        self.cap, self.group, self.depth = int(cap_rows), group, depth
        self.world = dist.get_world_size(group)
        self.device = device if device is not None else torch.device("cpu")
        self.cuda = self.device.type == "cuda"
        self.side = torch.cuda.Stream(device=self.device) if self.cuda else None
        rows = self.cap + 1
        self.bufs = [{"payload": torch.zeros((rows, 4), dtype=torch.int64, device=self.device),
                      "all": torch.zeros((self.world * rows, 4), dtype=torch.int64, device=self.device),
                      "scored": torch.cuda.Event() if self.cuda else None, "packed": torch.cuda.Event() if self.cuda else None,
                      "work": None, "unordered": False} for _ in range(depth)]
        self.i = 0

    def _all_gather(self, b):
        """One all-gather of b["payload"] into b["all"].  nccl (= RCCL): asynchronous, device to device.  gloo has no all-gather of
        device tensors: the payload is staged through host memory, synchronously (how a one-GPU box runs the N-rank bench path: every
        rank on the same device, tests/test_gpu_multi.py; never the driver's path)."""
        if self.cuda and dist.get_backend(self.group) == "gloo":
            torch.cuda.current_stream().synchronize()
            host_all = torch.empty(b["all"].shape, dtype=b["all"].dtype)
            dist.all_gather_into_tensor(host_all, b["payload"].cpu(), group=self.group)
            b["all"].copy_(host_all)
            torch.cuda.current_stream().synchronize()
            return _DoneWork()
        return dist.all_gather_into_tensor(b["all"], b["payload"], group=self.group, async_op=True)

    def next_buffers(self):
        """The buffer set of the next batch; its payload may be written once the all-gather that last used it is done
        (on CUDA the current stream is made to wait for it, on CPU the call blocks)."""
        b = self.bufs[self.i % self.depth]
        self.i += 1
        if b["work"] is not None:
            b["work"].wait()
            b["work"] = None
        return b

    def submit(self, b):
        """Launch the all-gather of b["payload"] (written by work already enqueued on the current stream)."""
        if self.cuda:
            b["scored"].record(torch.cuda.current_stream())
            with torch.cuda.stream(self.side):
                self.side.wait_event(b["scored"])
                b["work"] = self._all_gather(b)
        else:
            b["work"] = self._all_gather(b)
        return b

    def collect(self, b):
        """Host side of one batch: (rows [sum k_r, 4] int64 ordered by global index, counts per rank)."""
        if b["work"] is not None:
            if self.cuda:
                with torch.cuda.stream(self.side):
                    b["work"].wait()
                self.side.synchronize()
            else:
                b["work"].wait()
        rows = self.cap + 1
        counts = [int(b["all"][r * rows, 0]) for r in range(self.world)]
        if max(counts) > self.cap:
            raise OverflowError(f"a rank produced {max(counts)} records, capacity is {self.cap}: rerun the batch with a larger cap_rows")
        out = torch.cat([b["all"][r * rows + 1: r * rows + 1 + counts[r]] for r in range(self.world)], dim=0)
        if b.get("unordered"):  # rows appended by the scoring kernel arrive in any order; they carry their index
            out = out[torch.argsort(out[:, 0])]
        return out, counts

    def finish(self):
        for b in self.bufs:
            if b["work"] is not None:
                if self.cuda:
                    with torch.cuda.stream(self.side):
                        b["work"].wait()
                else:
                    b["work"].wait()
                b["work"] = None
        if self.cuda:
            self.side.synchronize()
            torch.cuda.current_stream().synchronize()


class StreamedGather(PayloadGather):
    """PayloadGather fed by the device — what `bench.py --gpus N` runs: no host round trip per batch.

    score_step: `hc_score_pack_device` — the scoring kernel's row-appending twin writes the payload itself (rows
    unordered), then the all-gather on a side stream, overlapping the scoring kernel of the next batch.
    step: for results that exist already: `hc_compact_pack_device` (the library's own ordered selection + pack kernel, rows ordered) on the
    side stream; call before_write(results) before overwriting a results tensor that a step may still be reading.
    """

    def __init__(self, scorer, n_local, base_index, cap_rows, group=None, depth=2, rec_fmt=0):
        super().__init__(cap_rows, group, depth, torch.device("cuda", torch.cuda.current_device()))
        self.sc, self.n, self.base, self.fmt = scorer, int(n_local), int(base_index), int(rec_fmt)  # rec_fmt: records.REC_FULL / REC_COMPACT
        # zero-initialised: entries beyond the count of a batch are stale but always valid indices
        self.idx = torch.zeros(max(self.n, 1), dtype=torch.int32, device=self.device)
        self.count = torch.zeros(1, dtype=torch.int64, device=self.device)
        self.packed = {}  # data_ptr of a results tensor -> event: its rows have been packed, it may be overwritten

    def score_step(self, d_in_ptr, d_results):
        """Score this rank's n candidate records (device pointer) into d_results and collect the batch."""
        b = self.next_buffers()
        b["unordered"] = self.sc.score_pack_device(d_in_ptr, self.n, d_results.data_ptr(), self.cap, self.base, b["payload"].data_ptr(),
                                                   torch.cuda.current_stream().cuda_stream, self.fmt)
        return self.submit(b)

    def before_write(self, d_results):
        """The current stream waits until the batch that last used `d_results` has been read out of it."""
        ev = self.packed.get(d_results.data_ptr())
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)

    def step(self, d_results):
        """Collect a batch whose n result records were written by work already enqueued on the current stream."""
        b = self.next_buffers()
        b["unordered"] = False
        b["scored"].record(torch.cuda.current_stream())
        with torch.cuda.stream(self.side):
            self.side.wait_event(b["scored"])
            # (self.idx / self.count are shared: consecutive batches are ordered on the side stream)
            self.sc.compact_pack_device(d_results.data_ptr(), self.n, self.idx.data_ptr(), self.count.data_ptr(), self.cap, self.base,
                                        b["payload"].data_ptr(), self.side.cuda_stream)
            b["packed"].record(self.side)
            self.packed[d_results.data_ptr()] = b["packed"]
            b["work"] = self._all_gather(b)
        return b
